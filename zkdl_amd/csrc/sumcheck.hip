// zkdl_amd/csrc/sumcheck.hip -- inner-product / Hadamard / binary sumcheck provers for gfx950
// (reference: proof.cu:55-200).
//
// The reference runs, per round, a step kernel that materialises three coefficient tensors, then
// reduces each (sum, or a full log-depth MLE chain of launches), then folds the operand tables with
// two more launches (proof.cu:72-96,110-137,165-188): ~(6 + 3 log n) launches, ~6x the table in
// HBM traffic, a malloc/free pair per temporary and a device sync after every launch.  Here:
//
//  * ONE fused kernel per "big" round reads the live table(s) once and (a) accumulates the three
//    round coefficients -- plain sums for the inner-product variant, eq(u[r+1:], g)-weighted sums
//    for the Hadamard / binary variants -- through a wave64 shuffle ladder + one LDS stage into
//    per-workgroup partials, (b) writes the folded table(s), and (c) in workgroup 0, reduces the
//    PREVIOUS round's partials into the transcript (the kernel boundary is the grid-wide barrier,
//    so no atomics or flags are needed).
//  * Once the table fits LDS (<= 1024 elements) ONE single-workgroup kernel finishes all remaining
//    rounds with the tables resident in LDS (k_sc_tail).
//  * The eq weights of round r factor as E_r[g] = A_r[g mod 2^c_r] * B[g >> c_r] with a FIXED split
//    of the challenge vector: B = eq(u[s:]) never changes and A_r = eq(u[r+1:s)) shrinks by one
//    variable per round; every A_r and B is produced by one prep launch (k_sc_prep).  The tail
//    kernel keeps its own weight table in LDS and halves it each round (E'[g] = E[2g] + E[2g+1]).
//
// A length-2^24 binary sumcheck is 1 + 14 + 1 launches instead of the reference's ~2000.  Fr results
// are canonical, so this re-association is bit-identical to the reference's order.
#include <type_traits>
#include <cstdlib>
#include "ctx.hpp"
#include "dev_util.hpp"
#include "fs29.hpp"

namespace zk {

enum { SC_IP = 0, SC_HP = 1, SC_BIN = 2 };
static constexpr uint32_t SC_TAIL_N = 1024;   // table length handled by the LDS-resident tail kernel
static constexpr uint32_t SC_TAIL_LOG = 10;
static constexpr uint32_t SC_MAX_BLOCKS = 512;

__device__ __forceinline__ Fr eq_product(const Fr* __restrict__ u, uint32_t nvars, uint32_t j) {
  Fr one = fr_one(), acc = one;
  for (uint32_t b = 0; b < nvars; b++) {
    Fr ub = u[b];
    acc = fr_mul(acc, ((j >> b) & 1u) ? ub : fr_sub(one, ub));
  }
  return acc;
}

// Weight tables of all big rounds in one launch.  w = weight challenges (u), k variables, split s:
//   B[j]            = eq(w[s .. k), j)                      j < 2^(k-s)
//   A[aoff(r) + j]  = eq(w[r+1 .. s), j)                    j < 2^(s-1-r),  r = 0 .. s-1,  aoff(r) = 2^s - 2^(s-r)
__device__ __forceinline__ void k_sc_prep(uint32_t zk_nblk, const Fr* __restrict__ w, uint32_t k, uint32_t s, Fr* __restrict__ A, Fr* __restrict__ B) {
  const uint32_t nA = (1u << s) - 1u, nB = 1u << (k - s);
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nA + nB; idx += zk_nblk * blockDim.x) {
    if (idx < nA) {
      // level r holds 2^(s-1-r) entries at offset 2^s - 2^(s-r):  2^(s-r-1) < 2^s - idx <= 2^(s-r)
      uint32_t rem = (1u << s) - idx;  // >= 2
      uint32_t r = s - (32u - __clz(rem - 1u));
      uint32_t j = idx - ((1u << s) - (1u << (s - r)));
      A[idx] = eq_product(w + r + 1, s - 1 - r, j);
    } else {
      uint32_t j = idx - nA;
      B[j] = eq_product(w + s, k - s, j);
    }
  }
}

// R consecutive big sumcheck rounds in one launch (R = 1, 2 or 3).  All challenges are known up front, so
// a lane can take 2^R consecutive elements, emit the coefficient terms of round r0 for its 2^(R-1) pairs,
// fold them, emit round r0+1 for the 2^(R-2) folded pairs, ... entirely in registers, and write only the
// table after R folds: R times fewer launches and (1 + 2^-R) instead of R*1.5 table passes over HBM.
// Missing elements (ragged sizes) read as zero, which reproduces Fr_me_step's odd-tail rule
// a0 - x*a0 = a0 + x*(0 - a0) and Fr_ip_sc_step's zero padding (proof.cu:61-64) exactly.
//   fold[r0 + j]  : fold challenge of round r0 + j (device memory)
//   Aj / cj       : low weight table and its width for round r0 + j ; B the fixed high table
//   partial[blk*3R + 3j + q], j < R, q < 3 ; prev_* : the previous launch's partials (3*prev_R per block)
// (the multi-round form and the 32-bit-limb single round never beat k_sc_round29 on MI355X: they are compiled into the EXPERIMENTS build only)
#ifdef ZKDL_EXPERIMENTS
static uint32_t sc_multi_threshold(int R) {
  static const uint32_t t3 = [] { const char* e = getenv("ZKDL_SC_R3_MIN"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 0xffffffffu; }();
  static const uint32_t t2 = [] { const char* e = getenv("ZKDL_SC_R2_MIN"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 0xffffffffu; }();
  return R == 3 ? t3 : t2;
}
#endif

struct ScLevels {
  const Fr* A[3];
  uint32_t c[3];
};
template <int MODE, int R>
__device__ __forceinline__ void k_sc_round(uint32_t zk_nblk, const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ a_out, Fr* __restrict__ b_out,
                                                   const Fr* __restrict__ fold, ScLevels lv, const Fr* __restrict__ B, uint32_t in_size, uint32_t out_size,
                                                   Fr* __restrict__ partial, const Fr* __restrict__ prev_partial, uint32_t prev_blocks, uint32_t prev_cols,
                                                   Fr* __restrict__ prev_dst) {
  __shared__ Fr lds[12];
  constexpr int NE = 1 << R;
  Fr x[R];
#pragma clang loop unroll(full)
  for (int j = 0; j < R; j++) x[j] = fold[j];
  Fr acc[3 * R];
#pragma clang loop unroll(full)
  for (int i = 0; i < 3 * R; i++) acc[i] = fr_zero();
  // Factored weights: the eq weight of pair h is A[h mod 2^c] * B[h >> c].  When the pairs divide evenly into
  // contiguous per-workgroup ranges that do not straddle a multiple of 2^c (the first big rounds of every
  // power-of-two table: most of the work), B is the same for the whole workgroup: lanes weight by A only and
  // lane 0 multiplies the three workgroup sums by B once -- one product less per pair.
  uint32_t P = 0;
  bool fact = false;
  if (R == 1 && MODE != SC_IP) {
    P = out_size / zk_nblk;
    fact = lv.c[2] != 0xffu && P * zk_nblk == out_size && P >= blockDim.x && P % blockDim.x == 0 && lv.c[0] < 32 && P <= (1u << lv.c[0]) && ((1u << lv.c[0]) % P) == 0;
  }
  // (two instantiations of the loop: a runtime select would let the compiler evaluate the product speculatively)
  auto sweep = [&](auto FACT) {
  const uint32_t g_first = FACT.value ? blockIdx.x * P + threadIdx.x : blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g_end = FACT.value ? (blockIdx.x + 1) * P : out_size;
  const uint32_t g_step = FACT.value ? blockDim.x : zk_nblk * blockDim.x;
  for (uint32_t g = g_first; g < g_end; g += g_step) {
    Fr va[NE], vb[MODE == SC_BIN ? 1 : NE];
    const uint64_t base = (uint64_t)g * NE;
#pragma clang loop unroll(full)
    for (int e = 0; e < NE; e++) {
      va[e] = base + e < in_size ? a[base + e] : fr_zero();
      if (MODE != SC_BIN) vb[e] = base + e < in_size ? b[base + e] : fr_zero();
    }
#pragma clang loop unroll(full)
    for (int j = 0; j < R; j++) {
      const int np = NE >> (j + 1);  // pairs of this lane in round r0 + j
      const uint32_t mask = (1u << lv.c[j]) - 1u;
#pragma clang loop unroll(full)
      for (int i = 0; i < np; i++) {
        const uint32_t h = g * np + i;  // pair index within the round
        Fr a0 = va[2 * i], a1 = va[2 * i + 1];
        Fr da = fr_sub(a1, a0);
        Fr c0, c1, c2;
        // weighted coefficients w*c_q with the weight folded into a0 and da first: 5 (binary) / 6 (Hadamard)
        // products instead of 3 + 3 / 4 + 3
        if (MODE == SC_BIN) {
          Fr w = lv.A[j][h & mask];
          if constexpr (!decltype(FACT)::value) w = fr_mul(w, B[h >> lv.c[j]]);
          const Fr a0w = fr_mul(a0, w), daw = fr_mul(da, w);
          c0 = fr_sub(fr_mul(a0w, a0), a0w);            // w (a0^2 - a0)
          c1 = fr_sub(fr_dbl(fr_mul(a0w, da)), daw);    // w (2 a0 da - da)
          c2 = fr_mul(daw, da);                         // w da^2
        } else {
          Fr b0 = vb[2 * i], b1 = vb[2 * i + 1];
          Fr db = fr_sub(b1, b0);
          if (MODE == SC_HP) {
            Fr w = lv.A[j][h & mask];
            if constexpr (!decltype(FACT)::value) w = fr_mul(w, B[h >> lv.c[j]]);
            const Fr a0w = fr_mul(a0, w), daw = fr_mul(da, w);
            c0 = fr_mul(a0w, b0);
            c2 = fr_mul(daw, db);
            c1 = fr_sub(fr_sub(fr_mul(fr_add(a0w, daw), b1), c0), c2);  // a0w db + daw b0 = (a0w + daw) b1 - c0 - c2
          } else {
            c0 = fr_mul(a0, b0);
            c2 = fr_mul(da, db);
            c1 = fr_sub(fr_sub(fr_mul(a1, b1), c0), c2);  // a0 db + b0 da = a1 b1 - c0 - c2
          }
          vb[i] = fr_add(b0, fr_mul(x[j], db));
        }
        va[i] = fr_add(a0, fr_mul(x[j], da));
        acc[3 * j + 0] = fr_add(acc[3 * j + 0], c0);
        acc[3 * j + 1] = fr_add(acc[3 * j + 1], c1);
        acc[3 * j + 2] = fr_add(acc[3 * j + 2], c2);
      }
    }
    a_out[g] = va[0];
    if (MODE != SC_BIN) b_out[g] = vb[0];
  }
  };
  if (fact) sweep(std::true_type{});
  else sweep(std::false_type{});
#pragma clang loop unroll(full)
  for (int j = 0; j < R; j++) {
    block_sum3_fr(acc[3 * j], acc[3 * j + 1], acc[3 * j + 2], lds);
    if (threadIdx.x == 0) {
#pragma clang loop unroll(full)
      for (int q = 0; q < 3; q++) {
        Fr t = acc[3 * j + q];
        if (fact) t = fr_mul(t, B[(blockIdx.x * P) >> lv.c[0]]);
        partial[(uint64_t)blockIdx.x * (3 * R) + 3 * j + q] = t;
      }
    }
  }
  // workgroup 0 also closes the previous launch: its partials were completed by the kernel boundary
  if (blockIdx.x == 0 && prev_blocks) {
    for (uint32_t q = 0; q < prev_cols; q++) {
      Fr t = fr_zero();
      for (uint32_t i = threadIdx.x; i < prev_blocks; i += blockDim.x) t = fr_add(t, prev_partial[(uint64_t)i * prev_cols + q]);
      t = block_sum_fr(t, lds);
      if (threadIdx.x == 0) prev_dst[q] = t;
    }
  }
}

// ---- One big round in the 29-bit-limb arithmetic of fs29.hpp: the same coefficients, folded tables and partials layout as
// k_sc_round<MODE, 1>, with
//   * every product a single-instruction-MAC column scan (fs_mul) and every sum / difference carry-free,
//   * the three coefficient products of a pair accumulated UNREDUCED (fs_mul_raw_acc: 81 MACs instead of a Montgomery
//     product) into 18 columns per coefficient per lane; columns are added across the wave by shuffles and across the
//     workgroup through LDS as plain integers, and lanes 0..2 turn the three totals into canonical field elements,
//   * SC29_PPT pairs per lane instead of one or two, so that the per-workgroup epilogue is amortised.
// The coefficients are algebraically the reference's (proof.cu:55-70,110-137,165-188) and every value written is canonical,
// so the transcript is bit-identical.  Bounds (units of r; "n" = normalised): table entries and weights < 1; differences
// x - y + 2r < 3; fs_mul(A, B) < 0.453 A B + 1.
//   binary:    c0 = a0w (a0 - 1),  c1 = daw (2 a0 - 1),  c2 = daw da            with a0w = a0 w < 1.7, daw = da w < 3
//   Hadamard:  c0 = a0w b0,  c2 = daw db,  c1 = (a0w + daw) b1 - c0 - c2
//   inner:     c0 = a0 b0,   c2 = da db,   c1 = a1 b1 - c0 - c2
static constexpr uint32_t SC29_PPT = 8;
template <int MODE>
__device__ __forceinline__ void k_sc_round29(uint32_t zk_nblk, const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ a_out, Fr* __restrict__ b_out,
                                                     const Fr* __restrict__ fold, ScLevels lv, const Fr* __restrict__ B, uint32_t in_size, uint32_t out_size,
                                                     Fr* __restrict__ partial, const Fr* __restrict__ prev_partial, uint32_t prev_blocks, uint32_t prev_cols,
                                                     Fr* __restrict__ prev_dst) {
  __shared__ uint32_t quads[64 * 54];
  __shared__ uint64_t tot[54];
  __shared__ Fr Y[3], cres[9];
  __shared__ Fr lds[12];
  const Fs X = fs_unpack(fold[0]);
  Fs ONE;
#pragma unroll
  for (int i = 0; i < 9; i++) ONE.v[i] = fs29::ONE(i);
  uint64_t acc64[3][18];  // 64-bit columns across the lane's pairs (fs_mul_raw_acc64): cut back to 29 bits every fourth pair
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int k = 0; k < 18; k++) acc64[q][k] = 0;
  // factored weights: see k_sc_round
  uint32_t P = 0;
  bool fact = false;
  if (MODE != SC_IP) {
    P = out_size / zk_nblk;
    fact = lv.c[2] != 0xffu && P * zk_nblk == out_size && P >= blockDim.x && P % blockDim.x == 0 && lv.c[0] < 32 && P <= (1u << lv.c[0]) && ((1u << lv.c[0]) % P) == 0;
  }
  const uint32_t mask = (1u << lv.c[0]) - 1u;
  const Fr zero = fr_zero();
  // the multipliers of the epilogue, computed while the other lanes already sweep: 2^(232 j) mod r, times the
  // workgroup's high weight when the weights factor (then the lanes weight by the low table only)
  if (threadIdx.x < 3) {
    Fr kj = fr_zero();
    if (threadIdx.x == 0) kj.v[0] = 1u;
    if (threadIdx.x == 1) kj.v[7] = 0x100u;
    if (threadIdx.x == 2) {
      constexpr uint32_t k2[8] = {0x528ebd8du, 0x6ebc8427u, 0x2e5747bau, 0x8a38d0a4u, 0xb00d5ba7u, 0xf98b22b4u, 0x247c4690u, 0x37402f0eu};  // 2^464 mod r
#pragma unroll
      for (int i = 0; i < 8; i++) kj.v[i] = k2[i];
    }
    if (fact) kj = fr_mul(kj, B[(blockIdx.x * P) >> lv.c[0]]);
    Y[threadIdx.x] = kj;
  }
  auto sweep = [&](auto FACT) {
    const uint32_t g_first = FACT.value ? blockIdx.x * P + threadIdx.x : blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g_end = FACT.value ? (blockIdx.x + 1) * P : out_size;
    const uint32_t g_step = FACT.value ? blockDim.x : zk_nblk * blockDim.x;
    uint32_t it = 0;
    for (uint32_t g = g_first; g < g_end; g += g_step) {
      const uint64_t e0 = 2ull * g, e1 = e0 + 1;
      const Fs A0 = fs_unpack(e0 < in_size ? a[e0] : zero), A1 = fs_unpack(e1 < in_size ? a[e1] : zero);
      const Fs DA = fs_norm(fs_sub<2>(A1, A0));  // < 3
      Fs W;
      if (MODE != SC_IP) {
        W = fs_unpack(lv.A[0][g & mask]);
        if constexpr (!decltype(FACT)::value) W = fs_mul(W, fs_unpack(B[g >> lv.c[0]]));  // < 1.5
      }
      if (MODE == SC_BIN) {
        const Fs A0W = fs_mul(A0, W), DAW = fs_mul(DA, W);                 // < 1.7, < 3
        fs_mul_raw_acc64(acc64[0], A0W, fs_norm(fs_sub<2>(A0, ONE)));           // a0 - 1: < 3
        fs_mul_raw_acc64(acc64[1], DAW, fs_norm(fs_sub<2>(fs_add(A0, A0), ONE)));  // 2 a0 - 1: < 4
        fs_mul_raw_acc64(acc64[2], DAW, DA);
      } else {
        const Fs B0 = fs_unpack(e0 < in_size ? b[e0] : zero), B1 = fs_unpack(e1 < in_size ? b[e1] : zero);
        const Fs DB = fs_norm(fs_sub<2>(B1, B0));
        if (MODE == SC_HP) {
          const Fs A0W = fs_mul(A0, W), DAW = fs_mul(DA, W);
          fs_mul_raw_acc64(acc64[0], A0W, B0);
          fs_mul_raw_acc64(acc64[1], fs_norm(fs_add(A0W, DAW)), B1);  // < 4.7
          fs_mul_raw_acc64(acc64[2], DAW, DB);
        } else {
          fs_mul_raw_acc64(acc64[0], A0, B0);
          fs_mul_raw_acc64(acc64[1], A1, B1);
          fs_mul_raw_acc64(acc64[2], DA, DB);
        }
        b_out[g] = fs_to_fr(fs_norm(fs_add(B0, fs_mul(X, DB))));  // < 1 + 2.4
      }
      a_out[g] = fs_to_fr(fs_norm(fs_add(A0, fs_mul(X, DA))));
      if ((++it & 3u) == 0) {
#pragma unroll
        for (int q = 0; q < 3; q++) fs_cols_norm64(acc64[q]);
      }
    }
  };
  if (fact) sweep(std::true_type{});
  else sweep(std::false_type{});
  // ---- the three column sets, lane -> workgroup, as plain integers: two DPP steps inside each quad (columns < 2^29 after
  // the carry, so a quad's sum fits 32 bits), one lane per quad to LDS, 54 lanes add a column each over the 64 quads.
  uint32_t acc[3][18];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    fs_cols_norm64(acc64[q]);
#pragma unroll
    for (int k = 0; k < 18; k++) acc[q][k] = (uint32_t)acc64[q][k];  // < 2^29 each (the top one: the whole sum / 2^493 < 2^25)
  }
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int k = 0; k < 18; k++) {
      uint32_t v = acc[q][k];
      v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
      v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
      acc[q][k] = v;
    }
  if ((threadIdx.x & 3u) == 0) {
    uint32_t* dst = quads + (threadIdx.x >> 2) * 54;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int k = 0; k < 18; k++) dst[q * 18 + k] = acc[q][k];
  }
  __syncthreads();
  if (threadIdx.x < 54) {
    uint64_t t = 0;
    for (uint32_t i = 0; i < blockDim.x / 4; i++) t += quads[i * 54 + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  // nine lanes: (coefficient q, 232-bit chunk j) -> chunk * Y_j, Y_j = 2^(232 j) [* the workgroup's high weight] (fs_cols_to_fr)
  if (threadIdx.x < 9) {
    const uint32_t q = threadIdx.x / 3, j = threadIdx.x % 3;
    uint32_t limb[24];
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 24; k++) {
      uint64_t t = c + (k < 18 ? tot[q * 18 + k] : 0);
      limb[k] = (uint32_t)t & fs29::MASK;
      c = t >> 29;
    }
    Fs ch;
#pragma unroll
    for (int i = 0; i < 8; i++) ch.v[i] = j == 0 ? limb[i] : (j == 1 ? limb[8 + i] : limb[16 + i]);
    ch.v[8] = 0;
    uint32_t top;
    cres[threadIdx.x] = fr_mul(fs_pack(ch, top), Y[j]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Fr t0 = fr_add(fr_add(cres[0], cres[1]), cres[2]);
    Fr t1 = fr_add(fr_add(cres[3], cres[4]), cres[5]);
    Fr t2 = fr_add(fr_add(cres[6], cres[7]), cres[8]);
    if (MODE != SC_BIN) t1 = fr_sub(fr_sub(t1, t0), t2);
    partial[(uint64_t)blockIdx.x * 3 + 0] = t0;
    partial[(uint64_t)blockIdx.x * 3 + 1] = t1;
    partial[(uint64_t)blockIdx.x * 3 + 2] = t2;
  }
  // workgroup 0 also closes the previous launch: its partials were completed by the kernel boundary
  if (blockIdx.x == 0 && prev_blocks) {
    __syncthreads();
    for (uint32_t q = 0; q < prev_cols; q++) {
      Fr t = fr_zero();
      for (uint32_t i = threadIdx.x; i < prev_blocks; i += blockDim.x) t = fr_add(t, prev_partial[(uint64_t)i * prev_cols + q]);
      t = block_sum_fr(t, lds);
      if (threadIdx.x == 0) prev_dst[q] = t;
    }
  }
}

// Round 0 of a Hadamard sumcheck whose second operand is a 0/1 tensor given as one BYTE per element (zkReLU's sign tensor:
// sign_packed): with b in {0, 1} four of the round's seven products are selects --
//     c0 = b0 ? a0 w : 0,   (a0 w + da w) b1 = b1 ? a1 w : 0,   c2 = da w (b1 - b0) = 0 or +-da w,   b' = b0 + x (b1 - b0) in {0, x, 1 - x, 1}
// -- and the operand is read as 1 byte instead of 32.  Same coefficients, same folded tables (now as field elements) as k_sc_round<SC_HP, 1>.
__device__ __forceinline__ void k_sc_round_hp0(uint32_t zk_nblk, const Fr* __restrict__ a, const uint8_t* __restrict__ bb, Fr* __restrict__ a_out, Fr* __restrict__ b_out,
                                                       const Fr* __restrict__ fold, ScLevels lv, const Fr* __restrict__ B, uint32_t in_size, uint32_t out_size,
                                                       Fr* __restrict__ partial) {
  __shared__ Fr lds[12];
  const Fr x = fold[0], one = fr_one(), zero = fr_zero();
  const Fr nx = fr_sub(one, x);
  Fr acc0 = zero, acc1 = zero, acc2 = zero;
  const uint32_t P = out_size / zk_nblk;
  const bool fact = lv.c[2] != 0xffu && P * zk_nblk == out_size && P >= blockDim.x && P % blockDim.x == 0 && lv.c[0] < 32 && P <= (1u << lv.c[0]) && ((1u << lv.c[0]) % P) == 0;
  const uint32_t mask = (1u << lv.c[0]) - 1u;
  auto sweep = [&](auto FACT) {
    const uint32_t g_first = FACT.value ? blockIdx.x * P + threadIdx.x : blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g_end = FACT.value ? (blockIdx.x + 1) * P : out_size;
    const uint32_t g_step = FACT.value ? blockDim.x : zk_nblk * blockDim.x;
    for (uint32_t g = g_first; g < g_end; g += g_step) {
      const uint64_t e0 = 2ull * g, e1 = e0 + 1;
      const Fr a0 = e0 < in_size ? a[e0] : zero, a1 = e1 < in_size ? a[e1] : zero;
      const bool b0 = e0 < in_size && bb[e0] != 0, b1 = e1 < in_size && bb[e1] != 0;
      Fr w = lv.A[0][g & mask];
      if constexpr (!decltype(FACT)::value) w = fr_mul(w, B[g >> lv.c[0]]);
      const Fr da = fr_sub(a1, a0);
      const Fr a0w = fr_mul(a0, w), daw = fr_mul(da, w);
      const Fr c0 = b0 ? a0w : zero;
      const Fr t = b1 ? fr_add(a0w, daw) : zero;
      const Fr c2 = b0 == b1 ? zero : (b1 ? daw : fr_neg(daw));
      acc0 = fr_add(acc0, c0);
      acc1 = fr_add(acc1, fr_sub(fr_sub(t, c0), c2));
      acc2 = fr_add(acc2, c2);
      a_out[g] = fr_add(a0, fr_mul(x, da));
      b_out[g] = b0 ? (b1 ? one : nx) : (b1 ? x : zero);
    }
  };
  if (fact) sweep(std::true_type{});
  else sweep(std::false_type{});
  block_sum3_fr(acc0, acc1, acc2, lds);
  if (threadIdx.x == 0) {
    Fr t0 = acc0, t1 = acc1, t2 = acc2;
    if (fact) {
      const Fr bw = B[(blockIdx.x * P) >> lv.c[0]];
      t0 = fr_mul(t0, bw);
      t1 = fr_mul(t1, bw);
      t2 = fr_mul(t2, bw);
    }
    partial[(uint64_t)blockIdx.x * 3 + 0] = t0;
    partial[(uint64_t)blockIdx.x * 3 + 1] = t1;
    partial[(uint64_t)blockIdx.x * 3 + 2] = t2;
  }
}

// All remaining rounds with the table(s) resident in LDS.  n <= SC_TAIL_N, rounds r0 .. k-1.
// fold = fold challenges (all k), w = weight challenges (all k) or nullptr for SC_IP.
// Also closes the last big round (prev_*), and writes the final a(0) [, b(0)].
template <int MODE>
__device__ __forceinline__ void k_sc_tail(uint32_t zk_nblk, const Fr* __restrict__ a_in, const Fr* __restrict__ b_in, uint32_t n, const Fr* __restrict__ fold,
                                                  const Fr* __restrict__ w, uint32_t k, uint32_t r0, Fr* __restrict__ proof,
                                                  const Fr* __restrict__ prev_partial, uint32_t prev_blocks, uint32_t prev_cols, Fr* __restrict__ prev_dst) {
  extern __shared__ Fr sm[];
  Fr* ta = sm;                                              // SC_TAIL_N
  Fr* tb = sm + SC_TAIL_N;                                  // SC_TAIL_N (unused for SC_BIN)
  Fr* E = sm + (MODE == SC_BIN ? SC_TAIL_N : 2 * SC_TAIL_N);  // SC_TAIL_N / 2 weights
  Fr* red = E + SC_TAIL_N / 2;                              // 24 reduction slots (3 per wave)
  const uint32_t tid = threadIdx.x;
  if (prev_blocks) {
    for (uint32_t q = 0; q < prev_cols; q++) {
      Fr acc = fr_zero();
      for (uint32_t i = tid; i < prev_blocks; i += blockDim.x) acc = fr_add(acc, prev_partial[(uint64_t)i * prev_cols + q]);
      acc = block_sum_fr(acc, red);
      if (tid == 0) prev_dst[q] = acc;
    }
  }
  for (uint32_t i = tid; i < n; i += blockDim.x) {
    ta[i] = a_in[i];
    if (MODE != SC_BIN) tb[i] = b_in[i];
  }
  uint32_t sz = n;
  if (MODE != SC_IP && r0 < k) {
    // weights of round r0 over the remaining variables w[r0+1 .. k)
    const uint32_t nv = k - r0 - 1;
    for (uint32_t j = tid; j < (1u << nv); j += blockDim.x) E[j] = eq_product(w + r0 + 1, nv, j);
  }
  __syncthreads();
  for (uint32_t r = r0; r < k; r++) {
    const uint32_t osz = (sz + 1) / 2;
    const Fr x = fold[r];
    Fr s0 = fr_zero(), s1 = fr_zero(), s2 = fr_zero();
    // with <= 512 outputs every lane owns at most one pair: read, barrier, then fold in place
    const uint32_t g = tid;
    Fr na = fr_zero(), nb = fr_zero();
    const bool act = g < osz;
    if (act) {
      const uint32_t g0 = 2 * g, g1 = 2 * g + 1;
      Fr a0 = ta[g0];
      Fr a1 = g1 < sz ? ta[g1] : fr_zero();
      Fr da = fr_sub(a1, a0);
      Fr c0, c1, c2;
      if (MODE == SC_BIN) {
        c0 = fr_sub(fr_sqr(a0), a0);
        c1 = fr_sub(fr_mul(fr_dbl(a0), da), da);
        c2 = fr_sqr(da);
      } else {
        Fr b0 = tb[g0];
        Fr b1 = g1 < sz ? tb[g1] : fr_zero();
        Fr db = fr_sub(b1, b0);
        c0 = fr_mul(a0, b0);
        c2 = fr_mul(da, db);
        c1 = fr_sub(fr_sub(fr_mul(a1, b1), c0), c2);
        nb = g1 < sz ? fr_add(b0, fr_mul(x, db)) : fr_sub(b0, fr_mul(x, b0));
      }
      na = g1 < sz ? fr_add(a0, fr_mul(x, da)) : fr_sub(a0, fr_mul(x, a0));
      if (MODE != SC_IP) {
        Fr wgt = E[g];
        c0 = fr_mul(c0, wgt);
        c1 = fr_mul(c1, wgt);
        c2 = fr_mul(c2, wgt);
      }
      s0 = c0; s1 = c1; s2 = c2;
    }
    // next round's weights: sum out the variable that pairs g: E'[j] = E[2j] + E[2j+1]
    Fr ne = fr_zero();
    const uint32_t nE = (MODE != SC_IP && r + 1 < k) ? (1u << (k - r - 2)) : 0u;
    if (g < nE) ne = fr_add(E[2 * g], E[2 * g + 1]);
    block_sum3_fr(s0, s1, s2, red);
    if (tid == 0) {
      proof[3 * r + 0] = s0;
      proof[3 * r + 1] = s1;
      proof[3 * r + 2] = s2;
    }
    __syncthreads();  // all reads of ta/tb/E of this round are done
    if (act) {
      ta[g] = na;
      if (MODE != SC_BIN) tb[g] = nb;
    }
    if (g < nE) E[g] = ne;
    __syncthreads();
    sz = osz;
  }
  if (tid == 0) {  // proof.cu:75-79,115-119,168-171
    proof[3 * k] = ta[0];
    if (MODE != SC_BIN) proof[3 * k + 1] = tb[0];
  }
}

static bool sc29_enabled() {
  static const bool off = experiment_on("ZKDL_SC_OLD");  // A/B switch (EXPERIMENTS build): the 32-bit-limb rounds
  return !off;
}
template <int MODE>
static int run_sumcheck(zk_ctx* ctx, const Fr* a_in, const Fr* b_in, uint32_t n, const Fr* fold_d, const Fr* weight_d, uint32_t k, Fr* proof_d, const uint8_t* b_bits = nullptr) {
  const bool two = (MODE != SC_BIN);
  // big rounds: while the live table is longer than SC_TAIL_N
  uint32_t nbig = 0;
  {
    uint32_t sz = n;
    while (sz > SC_TAIL_N && nbig < k) {
      sz = (sz + 1) / 2;
      nbig++;
    }
  }
  const uint32_t half = (n + 1) / 2;
  DBuf<Fr> abuf, bbuf, partial, tabA, tabB;
  uint32_t s = 0;
  if (nbig) {
    if (!abuf.alloc(ctx, (size_t)half + (half + 1) / 2 + 2) || (two && !bbuf.alloc(ctx, (size_t)half + (half + 1) / 2 + 2)) || !partial.alloc(ctx, 2 * 9 * SC_MAX_BLOCKS))
      return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: sumcheck tables");
    if (MODE != SC_IP) {
      // fixed split: every big round r (< nbig) needs c_r = s - 1 - r >= 0
      // (with the 29-bit-limb rounds a workgroup covers 2^11 pairs: a wide low table keeps the high weight constant per workgroup)
      s = nbig;
      static const bool wide_split = !experiment_on("ZKDL_SC29_NARROW");  // A/B switch (EXPERIMENTS build): keep the split at the number of big rounds (no factored weights then)
      if (sc29_enabled() && wide_split && k >= 6 && k - 6 > s) s = k - 6;
      if (s > k) s = k;
      if (!tabA.alloc(ctx, (size_t)1 << s) || !tabB.alloc(ctx, (size_t)1 << (k - s))) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: sumcheck weights");
      uint32_t tot = (1u << s) + (1u << (k - s));
      ZK_LAUNCHB(ctx, "sumcheck_prep", 32ull * tot, 256, k_sc_prep, dim3(stream_grid(tot, 256, 256)), 0, weight_d, k, s, tabA.get(), tabB.get());
    }
  }
  const Fr* a = a_in;
  const Fr* b = b_in;
  uint32_t sz = n;
  const uint32_t slot_elems = (half + 3) / 2 + 2;   // a launch of R >= 1 rounds writes <= ceil(n/2) elements; later ones less
  (void)slot_elems;
  Fr* a_slots[2] = {abuf.get(), abuf.get() ? abuf.get() + half + 1 : nullptr};
  Fr* b_slots[2] = {bbuf.get(), bbuf.get() ? bbuf.get() + half + 1 : nullptr};
  const char* tag = MODE == SC_IP ? "sumcheck_ip_round" : MODE == SC_HP ? "sumcheck_hp_round" : "sumcheck_bin_round";
  const Fr* prev_partial = nullptr;
  uint32_t prev_blocks = 0, prev_cols = 0;
  Fr* prev_dst = nullptr;
  uint32_t launch = 0;
  for (uint32_t r = 0; r < nbig;) {
    // Folding R rounds in one lane divides the lane count by 2^R and needs ~2x the registers: measured on
    // MI355X it never beat one round per launch on this workload (the rounds are VALU-bound, not HBM-bound),
    // so it is off unless ZKDL_SC_R2_MIN / ZKDL_SC_R3_MIN set a size threshold.
    uint32_t R = 1;
#ifdef ZKDL_EXPERIMENTS
    if (sz >= sc_multi_threshold(3)) R = 3;
    else if (sz >= sc_multi_threshold(2)) R = 2;
    if (R > nbig - r) R = nbig - r;
#endif
    uint32_t osz = sz;
    for (uint32_t j = 0; j < R; j++) osz = (osz + 1) / 2;
    ScLevels lv;
    for (uint32_t j = 0; j < 3; j++) {
      lv.A[j] = nullptr;
      lv.c[j] = 0;
      if (MODE != SC_IP && j < R) {
        lv.c[j] = s - 1 - (r + j);
        lv.A[j] = tabA.get() + ((1u << s) - (1u << (s - (r + j))));
      }
    }
    static const bool no_fact = experiment_on("ZKDL_SC_NOFACT");  // A/B switch (EXPERIMENTS build) for the factored-weight path
    if (no_fact && R == 1) lv.c[2] = 0xffu;
    uint32_t grid = stream_grid(osz, 256, SC_MAX_BLOCKS);
    Fr* ao = a_slots[launch & 1];
    Fr* bo = b_slots[launch & 1];
    Fr* part = partial.get() + (launch & 1) * 9 * SC_MAX_BLOCKS;
    uint64_t bytes = (two ? 2ull : 1ull) * 32ull * ((uint64_t)sz + osz);
    static const bool no_hp0 = experiment_on("ZKDL_NO_HP0");  // A/B switch (EXPERIMENTS build)
    if (MODE == SC_HP && b_bits && r == 0 && !no_hp0) {  // 0/1 second operand, one byte per element: round 0 without its products by b
      ZK_LAUNCHB(ctx, tag, 32ull * ((uint64_t)sz + 2ull * osz) + sz, 256, k_sc_round_hp0, dim3(grid), 0, a, b_bits, ao, bo, fold_d, lv, (const Fr*)tabB.get(), sz, osz, part);
      prev_partial = part;
      prev_blocks = grid;
      prev_cols = 3;
      prev_dst = proof_d;
      a = ao;
      b = bo;
      sz = osz;
      r += 1;
      launch++;
      continue;
    }
#define ZK_SC_LAUNCH(RR)                                                                                                                                   \
  ZK_LAUNCHB(ctx, tag, bytes, 256, (k_sc_round<MODE, RR>), dim3(grid), 0, a, b, ao, bo, fold_d + r, lv, (const Fr*)tabB.get(), sz, osz, part, prev_partial, \
            prev_blocks, prev_cols, prev_dst)
#ifdef ZKDL_EXPERIMENTS
    if (R == 3) ZK_SC_LAUNCH(3);
    else if (R == 2) ZK_SC_LAUNCH(2);
    else if (!sc29_enabled()) ZK_SC_LAUNCH(1);
    else
#endif
    {
      // SC29_PPT pairs per lane, but at least 64 workgroups while each still gets a full 256 pairs (the factored weights
      // need 2^(k-s) = 64 of them)
      static const uint32_t ppt = [] { const char* e = experiment_env("ZKDL_SC29_PPT"); uint32_t v = e ? (uint32_t)strtoul(e, nullptr, 0) : SC29_PPT; return v ? v : 1u; }();
      uint32_t g29 = stream_grid((osz + ppt - 1) / ppt, 256, SC_MAX_BLOCKS);
      if (g29 < 64) g29 = osz / 256 >= 64 ? 64 : stream_grid(osz, 256, 64);
      grid = g29;
      ZK_LAUNCHB(ctx, tag, bytes, 256, (k_sc_round29<MODE>), dim3(grid), 0, a, b, ao, bo, fold_d + r, lv, (const Fr*)tabB.get(), sz, osz, part, prev_partial,
                 prev_blocks, prev_cols, prev_dst);
    }
#undef ZK_SC_LAUNCH
    prev_partial = part;
    prev_blocks = grid;
    prev_cols = 3 * R;
    prev_dst = proof_d + 3 * r;
    a = ao;
    b = bo;
    sz = osz;
    r += R;
    launch++;
  }
  size_t lds = ((MODE == SC_BIN ? 1 : 2) * SC_TAIL_N + SC_TAIL_N / 2 + 24) * sizeof(Fr);
  if (ctx->lds_first_use(zk::Kern<512, &k_sc_tail<MODE>>::desc())) {  // > 64 KiB of dynamic LDS needs the opt-in, per device
    zk::Kern<512, &k_sc_tail<MODE>>::set_max_lds(lds);
    (void)hipGetLastError();
  }
  const char* ttag = MODE == SC_IP ? "sumcheck_ip_tail" : MODE == SC_HP ? "sumcheck_hp_tail" : "sumcheck_bin_tail";
  ZK_LAUNCHB(ctx, ttag, (two ? 2ull : 1ull) * 32ull * sz, 512, k_sc_tail<MODE>, dim3(1), lds, a, b, sz, fold_d, weight_d, k, nbig, proof_d, prev_partial, prev_blocks, prev_cols, prev_dst);
  return ZK_OK;
}

// inner_product_sumcheck guards: proof.cu:98-108
int sumcheck_ip_dev(zk_ctx* ctx, const Fr* a, const Fr* b, uint32_t n, const Fr* u_d, uint32_t k, Fr* proof_d) {
  if (n == 0 || k > 31 || n <= ((1u << k) / 2) || n > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  return run_sumcheck<SC_IP>(ctx, a, b, n, u_d, nullptr, k, proof_d);
}
// hadamard_product_sumcheck guards: proof.cu:139-150
// b_bits (optional): b as one byte per element, for a 0/1-valued b (the values of b[] must then be 0 or the Montgomery one)
int sumcheck_hp_dev(zk_ctx* ctx, const Fr* a, const Fr* b, uint32_t n, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d, const uint8_t* b_bits) {
  if (n == 0 || k > 31 || n > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions 4");
  if (k > 0 && n <= (1u << (k - 1))) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions 3");
  return run_sumcheck<SC_HP>(ctx, a, b, n, v_d, u_d, k, proof_d, b_bits);
}
// binary_sumcheck guards: proof.cu:190-200
int sumcheck_bin_dev(zk_ctx* ctx, const Fr* a, uint32_t n, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d) {
  if (n == 0 || k > 31 || n <= ((1u << k) / 2) || n > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  return run_sumcheck<SC_BIN>(ctx, a, nullptr, n, v_d, u_d, k, proof_d);
}

}  // namespace zk

// tools/experiments/fma_probe.hip -- VERDICT r1 item 5: a double-precision-FMA Montgomery multiplier for Fp, measured against the
// 32-bit-limb integer one (field.hpp).  Same radix as the product path (R = 2^384), so the results are the same Montgomery forms and are
// checked limb for limb; the only difference is the limb split: 8 limbs of 48 bits held as doubles instead of 12 of 32 bits.
//
// With round-toward-zero, h = fma(a, b, C) with C = 2^100 has ulp 2^48 and equals C + floor(ab / 2^48) 2^48 exactly (ab < 2^96), so
// hi = h - C and lo = fma(a, b, -hi) split a 48x48-bit product exactly.  Chaining h through a column (h = fma(a, b, h)) accumulates the
// high parts for free, so a partial product costs: fma (high chain), sub (this product's high part), fma (low part), add (low sum) = 4
// instructions for 2304 bit^2, against mad + addc = 2 for 1024 bit^2 -- 12 % more product bits per instruction, before the per-column
// Montgomery step (two more exact splits) and the carries.  v_fma_f64 / v_add_f64 issue at 4.2-4.3 cycles, v_mad_u64_u32 at 4.6.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkdl_amd/csrc tools/experiments/fma_probe.hip -o /tmp/fma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "field.hpp"
using namespace zk;

struct Fd { double v[8]; };  // 8 x 48-bit limbs as doubles, value < p

__device__ __forceinline__ void set_rz() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3"); }  // FP64 round toward zero

struct FdConst { double p[8]; double pinv; };

__device__ __forceinline__ Fd fd_mul(const Fd& a, const Fd& b, const FdConst& K) {
  const double C = 0x1p100, T48 = 0x1p48, I48 = 0x1p-48, C52 = 0x1p100;  // C52: same trick to floor(x / 2^48) of x < 2^60 scaled
  double m[8];
  Fd r;
  double carry = 0.0;  // carry into the current column, in units of 1
#pragma unroll
  for (int k = 0; k < 15; k++) {
    double h = C, lo = carry;
#pragma unroll
    for (int i = (k < 8 ? 0 : k - 7); i <= (k < 8 ? k : 7); i++) {
      const double h2 = __builtin_fma(a.v[i], b.v[k - i], h);
      const double hi = h2 - h;
      lo += __builtin_fma(a.v[i], b.v[k - i], -hi);
      h = h2;
    }
#pragma unroll
    for (int i = (k < 8 ? 0 : k - 7); i <= (k < 8 ? k - 1 : 7); i++) {
      const double h2 = __builtin_fma(m[i], K.p[k - i], h);
      const double hi = h2 - h;
      lo += __builtin_fma(m[i], K.p[k - i], -hi);
      h = h2;
    }
    if (k < 8) {
      // m_k = (lo mod 2^48) * pinv mod 2^48 ; then add m_k * p_0: the column becomes a multiple of 2^48
      const double q = (__builtin_fma(lo, I48, 0x1p52) - 0x1p52);   // floor(lo / 2^48): lo < 2^53 -> lo/2^48 < 32, exact floor under RZ
      const double t = lo - q * T48;
      const double hh = __builtin_fma(t, K.pinv, C) - C;
      const double mk = __builtin_fma(t, K.pinv, -hh);
      m[k] = mk;
      const double h2 = __builtin_fma(mk, K.p[0], h);
      const double hi = h2 - h;
      lo += __builtin_fma(mk, K.p[0], -hi);
      h = h2;
      carry = lo * I48 + (h - C) * I48;   // exact: lo is a multiple of 2^48 now
    } else {
      const double q = (__builtin_fma(lo, I48, 0x1p52) - 0x1p52);
      r.v[k - 8] = lo - q * T48;
      carry = q + (h - C) * I48;
    }
  }
  r.v[7] = carry;
  (void)C52;
  // conditional subtraction of p (value < 2p): borrow chain in doubles
  double d[8], bw = 0.0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    double x = r.v[i] - K.p[i] - bw;
    bw = x < 0.0 ? 1.0 : 0.0;
    d[i] = x < 0.0 ? x + T48 : x;
  }
  if (bw == 0.0) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = d[i];
  }
  return r;
}

// ---- host conversions
static void to48(const uint32_t* w, double* out) {  // 12 x u32 -> 8 x 48-bit
  unsigned __int128 acc = 0;
  int bits = 0, wi = 0;
  for (int j = 0; j < 8; j++) {
    while (bits < 48 && wi < 12) { acc |= (unsigned __int128)w[wi++] << bits; bits += 32; }
    out[j] = (double)(uint64_t)(acc & (((unsigned __int128)1 << 48) - 1));
    acc >>= 48; bits -= 48;
  }
}
static void from48(const double* in, uint32_t* w) {
  unsigned __int128 acc = 0;
  int bits = 0, wi = 0;
  for (int j = 0; j < 8; j++) {
    acc |= (unsigned __int128)(uint64_t)in[j] << bits; bits += 48;
    while (bits >= 32 && wi < 12) { w[wi++] = (uint32_t)acc; acc >>= 32; bits -= 32; }
  }
}

__global__ void __launch_bounds__(256) k_int(const Fp* in, Fp* out, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  Fp a = in[i], b = in[i + 1];
  for (int it = 0; it < iters; it++) { a = fp_mul(a, b); b = fp_mul(b, a); }
  out[i] = fp_add(a, b);
}
__global__ void __launch_bounds__(256) k_fma(const Fd* in, Fd* out, FdConst K, int iters) {
  set_rz();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  Fd a = in[i], b = in[i + 1];
  for (int it = 0; it < iters; it++) { a = fd_mul(a, b, K); b = fd_mul(b, a, K); }
#pragma unroll
  for (int l = 0; l < 8; l++) a.v[l] += b.v[l];
  out[i] = a;
}
__global__ void k_check(const Fp* ai, const Fp* bi, const Fd* ad, const Fd* bd, Fp* oi, Fd* od, FdConst K, int n) {
  set_rz();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  oi[i] = fp_mul(ai[i], bi[i]);
  od[i] = fd_mul(ad[i], bd[i], K);
}

int main() {
  FdConst K;
  uint32_t p32[12];
  for (int i = 0; i < 12; i++) p32[i] = FpP::P(i);
  to48(p32, K.p);
  {  // pinv = -p^-1 mod 2^48 by Newton iteration on 64-bit integers
    uint64_t p0 = (uint64_t)K.p[0], x = 1;
    for (int it = 0; it < 6; it++) x *= 2 - p0 * x;
    K.pinv = (double)((0 - x) & ((1ull << 48) - 1));
  }
  // correctness: random canonical operands
  const int n = 4096;
  Fp *ha = new Fp[n], *hb = new Fp[n];
  Fd *hda = new Fd[n], *hdb = new Fd[n];
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  for (int i = 0; i < n; i++) {
    for (int l = 0; l < 12; l++) { ha[i].v[l] = rnd(); hb[i].v[l] = rnd(); }
    ha[i].v[11] %= 436277738u; hb[i].v[11] %= 436277738u;
    if (i == 0) memset(&ha[i], 0, sizeof(Fp));
    if (i == 1) { for (int l = 0; l < 12; l++) ha[i].v[l] = FpP::P(l); ha[i].v[0] -= 1; hb[i] = ha[i]; }   // (p-1)^2
    to48(ha[i].v, hda[i].v); to48(hb[i].v, hdb[i].v);
  }
  Fp *da, *db, *doi; Fd *dda, *ddb, *dod;
  hipMalloc(&da, n * sizeof(Fp)); hipMalloc(&db, n * sizeof(Fp)); hipMalloc(&doi, n * sizeof(Fp));
  hipMalloc(&dda, n * sizeof(Fd)); hipMalloc(&ddb, n * sizeof(Fd)); hipMalloc(&dod, n * sizeof(Fd));
  hipMemcpy(da, ha, n * sizeof(Fp), hipMemcpyHostToDevice); hipMemcpy(db, hb, n * sizeof(Fp), hipMemcpyHostToDevice);
  hipMemcpy(dda, hda, n * sizeof(Fd), hipMemcpyHostToDevice); hipMemcpy(ddb, hdb, n * sizeof(Fd), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, da, db, dda, ddb, doi, dod, K, n);
  Fp* ro = new Fp[n]; Fd* rd = new Fd[n];
  hipMemcpy(ro, doi, n * sizeof(Fp), hipMemcpyDeviceToHost); hipMemcpy(rd, dod, n * sizeof(Fd), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++) {
    uint32_t w[12];
    from48(rd[i].v, w);
    if (memcmp(w, ro[i].v, 48) != 0) { if (bad < 3) printf("mismatch at %d\n", i); bad++; }
  }
  printf("correctness: %d / %d products differ from the integer multiplier\n", bad, n);
  // throughput
  const int blocks = 4096, threads = 256, iters = 200, N = blocks * threads + 1;
  Fp *ip, *op; Fd *iq, *oq;
  hipMalloc(&ip, N * sizeof(Fp)); hipMalloc(&op, N * sizeof(Fp)); hipMalloc(&iq, N * sizeof(Fd)); hipMalloc(&oq, N * sizeof(Fd));
  hipMemset(ip, 0x11, N * sizeof(Fp));
  { Fd* t = new Fd[N]; for (int i = 0; i < N; i++) for (int l = 0; l < 8; l++) t[i].v[l] = (double)((0x123456789abull * (i + 3 * l + 1)) & ((1ull << 47) - 1)); hipMemcpy(iq, t, N * sizeof(Fd), hipMemcpyHostToDevice); delete[] t; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    float ms;
    hipEventRecord(e0); hipLaunchKernelGGL(k_int, dim3(blocks), dim3(threads), 0, 0, ip, op, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("u32 x 12 limbs (v_mad_u64_u32): %.3f ms  -> %.1f G products/s\n", ms, 2.0 * iters * blocks * threads / ms / 1e6);
    hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, iq, oq, K, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("f64 x 8 limbs (v_fma_f64)     : %.3f ms  -> %.1f G products/s\n", ms, 2.0 * iters * blocks * threads / ms / 1e6);
  }
  return bad ? 1 : 0;
}

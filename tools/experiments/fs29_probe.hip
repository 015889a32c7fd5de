// Device-vs-host check of the 29-bit-limb arithmetic (zkdl_amd/csrc/fs29.hpp): the per-pair sequence of k_sc_round29 on random
// Montgomery-form operands, computed by a kernel and by the same inline code on the CPU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkdl_amd/csrc -o /tmp/fs29_probe tools/experiments/fs29_probe.hip && /tmp/fs29_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fs29.hpp"
using namespace zk;

__host__ __device__ inline void pair_coeffs(const Fr* in, Fr* out) {
  Fs A0 = fs_unpack(in[0]), A1 = fs_unpack(in[1]), B0 = fs_unpack(in[2]), B1 = fs_unpack(in[3]);
  Fs W = fs_unpack(in[4]);
  W = fs_mul(W, fs_unpack(in[5]));
  Fs ONE;
  for (int i = 0; i < 9; i++) ONE.v[i] = fs29::ONE(i);
  Fs DA = fs_norm(fs_sub<2>(A1, A0)), DB = fs_norm(fs_sub<2>(B1, B0));
  Fs A0W = fs_mul(A0, W), DAW = fs_mul(DA, W);
  uint32_t acc[6][18];
  for (int q = 0; q < 6; q++)
    for (int k = 0; k < 18; k++) acc[q][k] = 0;
  fs_mul_raw_acc(acc[0], A0W, fs_norm(fs_sub<2>(A0, ONE)));
  fs_mul_raw_acc(acc[1], DAW, fs_norm(fs_sub<2>(fs_add(A0, A0), ONE)));
  fs_mul_raw_acc(acc[2], DAW, DA);
  fs_mul_raw_acc(acc[3], A0W, B0);
  fs_mul_raw_acc(acc[4], fs_norm(fs_add(A0W, DAW)), B1);
  fs_mul_raw_acc(acc[5], DAW, DB);
  for (int q = 0; q < 6; q++) {
    uint64_t cols[18];
    for (int k = 0; k < 18; k++) cols[k] = acc[q][k];
    out[q] = fs_cols_to_fr(cols);
  }
  out[6] = fs_to_fr(fs_norm(fs_add(A0, fs_mul(fs_unpack(in[5]), DA))));
  out[7] = fs_to_fr(A0W);
  out[8] = fs_to_fr(W);
  out[9] = fs_to_fr(fs_mul(A0, fs_unpack(in[4])));
  out[10] = fs_to_fr(DAW);
  Fs W2 = W;  // W's limbs, raw
  for (int i = 0; i < 8; i++) out[11].v[i] = W2.v[i];
  for (int i = 0; i < 8; i++) out[12].v[i] = A0W.v[i];
  out[13].v[0] = W2.v[8]; out[13].v[1] = A0W.v[8];
  for (int i = 2; i < 8; i++) out[13].v[i] = 0;
}
__global__ void k_probe(const Fr* in, Fr* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pair_coeffs(in + 6 * i, out + 14 * i);
}
int main() {
  const int n = 4096;
  std::vector<Fr> in(6 * n), dev(14 * n), host(14 * n);
  srand(7);
  for (auto& x : in) {
    for (int i = 0; i < 8; i++) x.v[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    x.v[7] &= 0x3fffffffu;  // < r
  }
  for (int i = 0; i < 8; i++) { in[4].v[i] = FrP::ONE(i); in[5].v[i] = FrP::ONE(i); in[10].v[i] = 0; }
  Fr *din, *dout;
  hipMalloc(&din, in.size() * sizeof(Fr));
  hipMalloc(&dout, dev.size() * sizeof(Fr));
  hipMemcpy(din, in.data(), in.size() * sizeof(Fr), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
  hipMemcpy(dev.data(), dout, dev.size() * sizeof(Fr), hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++) pair_coeffs(&in[6 * i], &host[14 * i]);
  int bad = 0;
  int per[14] = {0};
  for (int i = 0; i < 14 * n; i++)
    if (memcmp(&dev[i], &host[i], sizeof(Fr))) {
      per[i % 14]++;
      if (bad < 3 && i % 14 >= 11) {
        printf("pair %d output %d\n dev ", i / 14, i % 14);
        for (int j = 0; j < 8; j++) printf("%08x ", dev[i].v[j]);
        printf("\n host ");
        for (int j = 0; j < 8; j++) printf("%08x ", host[i].v[j]);
        printf("\n");
        bad++;
      }
    }
  for (int q = 0; q < 14; q++) printf("output %d: %d of %d differ\n", q, per[q], n);
  return bad != 0;
}

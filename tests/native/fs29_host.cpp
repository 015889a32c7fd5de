// Host-side driver for tests/test_fs29.py: the 29-bit-limb scalar-field arithmetic of zkdl_amd/csrc/fs29.hpp (plain C++,
// compiled for the CPU) on operands read from stdin; results in canonical form as hex.
//   mul a b            -> fs_to_fr(fs_mul(a, b))
//   lazy x y           -> fs_to_fr(fs_mul(x + y, norm(x - y + 2r)))
//   fold a0 a1 x       -> fs_to_fr(norm(a0 + fs_mul(x, norm(a1 - a0 + 2r))))
//   raw n a1 b1 ...    -> fs_cols_to_fr(sum_i a_i * b_i as raw columns), operands a_i + 2r - b_i and b_i to exercise lazy values
//   pack a             -> fs_pack(fs_unpack(a))
#include <cstdio>
#include <cstring>
#include <string>
#include <iostream>
#include "../../zkdl_amd/csrc/fs29.hpp"
using namespace zk;

static Fr parse(const std::string& h) {
  Fr r = fr_zero();
  std::string s = h;
  while (s.size() < 64) s = "0" + s;
  for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)strtoul(s.substr(64 - 8 * (i + 1), 8).c_str(), nullptr, 16);
  return r;
}
static void show(const Fr& a) {
  for (int i = 7; i >= 0; i--) printf("%08x", a.v[i]);
  printf("\n");
}
int main() {
  std::string op;
  while (std::cin >> op) {
    if (op == "mul") {
      std::string a, b; std::cin >> a >> b;
      show(fs_to_fr(fs_mul(fs_unpack(parse(a)), fs_unpack(parse(b)))));
    } else if (op == "lazy") {
      std::string a, b; std::cin >> a >> b;
      Fs x = fs_unpack(parse(a)), y = fs_unpack(parse(b));
      show(fs_to_fr(fs_mul(fs_add(x, y), fs_norm(fs_sub<2>(x, y)))));
    } else if (op == "fold") {
      std::string a, b, c; std::cin >> a >> b >> c;
      Fs a0 = fs_unpack(parse(a)), a1 = fs_unpack(parse(b)), x = fs_unpack(parse(c));
      show(fs_to_fr(fs_norm(fs_add(a0, fs_mul(x, fs_norm(fs_sub<2>(a1, a0)))))));
    } else if (op == "raw") {
      int n; std::cin >> n;
      uint32_t acc[18]; memset(acc, 0, sizeof acc);
      uint64_t cols[18]; memset(cols, 0, sizeof cols);
      for (int i = 0; i < n; i++) {
        std::string a, b; std::cin >> a >> b;
        Fs x = fs_unpack(parse(a)), y = fs_unpack(parse(b));
        fs_mul_raw_acc(acc, fs_norm(fs_sub<2>(x, y)), y);
        if (i % 7 == 6) fs_cols_norm(acc);  // seven products fit between two carries
        if ((i & 63) == 63) {  // what the lanes of a wave / the waves of a workgroup do: add columns as integers
          for (int k = 0; k < 18; k++) { cols[k] += acc[k]; acc[k] = 0; }
        }
      }
      for (int k = 0; k < 18; k++) cols[k] += acc[k];
      show(fs_cols_to_fr(cols));
    } else if (op == "raw64") {  // the 64-bit columns of k_sc_round29: six products between two carries, the lane's columns cut to 29 bits at the end
      int n; std::cin >> n;
      uint64_t acc[18]; memset(acc, 0, sizeof acc);
      uint64_t cols[18]; memset(cols, 0, sizeof cols);
      for (int i = 0; i < n; i++) {
        std::string a, b; std::cin >> a >> b;
        Fs x = fs_unpack(parse(a)), y = fs_unpack(parse(b));
        fs_mul_raw_acc64(acc, fs_norm(fs_sub<2>(x, y)), y);
        if (i % 6 == 5) fs_cols_norm64(acc);
        if ((i & 63) == 63) {
          fs_cols_norm64(acc);
          for (int k = 0; k < 18; k++) { cols[k] += (uint32_t)acc[k]; if (acc[k] >> 32) printf("wide!"); acc[k] = 0; }
        }
      }
      fs_cols_norm64(acc);
      for (int k = 0; k < 18; k++) { cols[k] += (uint32_t)acc[k]; if (acc[k] >> 32) printf("wide!"); }
      show(fs_cols_to_fr(cols));
    } else if (op == "pair") {  // one pair of a weighted round, as k_sc_round29 computes it: bin c0 c1 c2, hp c0 S c2
      std::string h[6];
      for (auto& x : h) std::cin >> x;
      Fs A0 = fs_unpack(parse(h[0])), A1 = fs_unpack(parse(h[1])), B0 = fs_unpack(parse(h[2])), B1 = fs_unpack(parse(h[3]));
      Fs W = fs_mul(fs_unpack(parse(h[4])), fs_unpack(parse(h[5])));
      Fs ONE;
      for (int i = 0; i < 9; i++) ONE.v[i] = fs29::ONE(i);
      Fs DA = fs_norm(fs_sub<2>(A1, A0)), DB = fs_norm(fs_sub<2>(B1, B0));
      Fs A0W = fs_mul(A0, W), DAW = fs_mul(DA, W);
      uint32_t acc[6][18];
      memset(acc, 0, sizeof acc);
      fs_mul_raw_acc(acc[0], A0W, fs_norm(fs_sub<2>(A0, ONE)));
      fs_mul_raw_acc(acc[1], DAW, fs_norm(fs_sub<2>(fs_add(A0, A0), ONE)));
      fs_mul_raw_acc(acc[2], DAW, DA);
      fs_mul_raw_acc(acc[3], A0W, B0);
      fs_mul_raw_acc(acc[4], fs_norm(fs_add(A0W, DAW)), B1);
      fs_mul_raw_acc(acc[5], DAW, DB);
      for (int q = 0; q < 6; q++) {
        uint64_t cols[18];
        for (int k = 0; k < 18; k++) cols[k] = acc[q][k];
        show(fs_cols_to_fr(cols));
      }
    } else if (op == "pack") {
      std::string a; std::cin >> a;
      uint32_t top;
      Fr r = fs_pack(fs_unpack(parse(a)), top);
      if (top) printf("top!");
      show(r);
    }
  }
  return 0;
}

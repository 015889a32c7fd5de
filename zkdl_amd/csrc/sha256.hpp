// zkdl_amd/csrc/sha256.hpp -- SHA-256 compression (FIPS 180-4) as plain __host__ __device__ code, and the round-by-round
// Fiat-Shamir step built on it: the prover's device kernels (fs_rounds.hip) and the verifier's host code (verify.hip) run the SAME
// function, so a challenge is reproducible bit for bit on both sides.
//
//   state_{r+1} = SHA-256( state_r (32 bytes) | c0 | c1 | c2 )          c_q: the round's three coefficients, 32 bytes each, as stored
//                                                                       (little-endian uint32 limbs of the Montgomery form)
//   x_r         = the 32 digest bytes read as 8 little-endian uint32 limbs, top limb mod 1944954707   (random_vec's distribution,
//                                                                       proof.cu:3-11; the same mapping as zk_fs_derive)
// The message is 128 bytes, i.e. exactly two blocks plus one padding block.
#pragma once
#include <stdint.h>

#include "field.hpp"

namespace zk {

ZK_HD uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
ZK_HD uint32_t sha_bswap(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }

ZK_HD void sha256_init(uint32_t h[8]) {
  h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
  h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
}

// one 64-byte block given as 16 big-endian words
ZK_HD void sha256_compress(uint32_t h[8], const uint32_t blk[16]) {
  const uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u,
      0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du,
      0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu,
      0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u,
      0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
  // fully unrolled on the device: w[] and K[] then live in registers and literals (indexed by a loop counter they are scratch and
  // constant memory, ~3x the time of a compression on the one lane that hashes between two sumcheck rounds)
  uint32_t w[64];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = blk[i];
#pragma unroll
  for (int i = 16; i < 64; i++) {
    const uint32_t s0 = sha_rotr(w[i - 15], 7) ^ sha_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    const uint32_t s1 = sha_rotr(w[i - 2], 17) ^ sha_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    const uint32_t S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
    const uint32_t S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// The running Fiat-Shamir state of one sumcheck: the 32 digest bytes, kept as the 8 words of the digest (big-endian words, the
// natural form of SHA-256's output).  Stored in device memory in a 32-byte slot of the challenge array.
struct FsState {
  uint32_t h[8];
};

// state <- SHA-256(state | c0 | c1 | c2); returns the round challenge
ZK_HD Fr fs_round_step(FsState& st, const Fr& c0, const Fr& c1, const Fr& c2) {
  uint32_t h[8];
  sha256_init(h);
  uint32_t blk[16];
  // block 0: state (already big-endian words) | c0 (byte stream of little-endian limbs = byte-swapped words)
  for (int i = 0; i < 8; i++) blk[i] = st.h[i];
  for (int i = 0; i < 8; i++) blk[8 + i] = sha_bswap(c0.v[i]);
  sha256_compress(h, blk);
  for (int i = 0; i < 8; i++) blk[i] = sha_bswap(c1.v[i]);
  for (int i = 0; i < 8; i++) blk[8 + i] = sha_bswap(c2.v[i]);
  sha256_compress(h, blk);
  blk[0] = 0x80000000u;  // padding: 0x80, zeros, the message length in bits (128 bytes = 1024)
  for (int i = 1; i < 15; i++) blk[i] = 0;
  blk[15] = 1024u;
  sha256_compress(h, blk);
  Fr x;
  for (int i = 0; i < 8; i++) {
    st.h[i] = h[i];
    x.v[i] = sha_bswap(h[i]);  // digest bytes read as little-endian limbs
  }
  x.v[7] %= 1944954707u;
  return x;
}

}  // namespace zk

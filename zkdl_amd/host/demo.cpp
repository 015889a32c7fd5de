// zkdl_amd/host/demo.cpp -- the reference driver (demo.cu) on the MI355X library.
//   ./demo traced_model.pt sample_input.pt [--seed S] [--dump DIR] [--fs | --sound] [--verify] [--repeat N] [--gpus N] [--dist] [--full-tables C]
// Same CLI, same inputs (TorchScript nn.Sequential of bias-free Linear + ReLU; a tensor saved as
// parameter "0" of a scripted empty module, model.py:4-8), same outputs: demo.out (Y_hat, one
// 0x%064x per line, demo.cu:119-122) and the three stdout lines (demo.cu:92,140-141).
// Differences: tensors may live on the CPU (the reference hard-fails, demo.cu:79,114); the contiguous
// weight tensor is kept alive while its pointer is used (the reference dangles, demo.cu:78).
// --dump DIR additionally writes what a checker needs to replay the proof: gens_<i>.bin (the randomised generators,
// Jacobian), challenges.bin (every random_vec draw, in order) and transcript.bin (zk_transcript_serialize).
// --fs derives the challenges of every prove call from SHA-256(model file | input file | commitments) (zk_fs_derive) instead of
// the seeded generator; --verify then checks the serialised proof call by call with zk_verify_fc / zk_verify_relu, reading
// only public data (generators, commitments, the input and output tensors), and prints "Verification: OK" or the failed check.
// --sound is --fs with every sumcheck ROUND challenge a hash of the rounds before it (zkFC::prove_fs / zkReLU::prove_fs -> zk_fc_prove_fs /
// zk_relu_prove_fs, the hash running on the device between a round's message and its fold): a prover is bound by each message before it learns the next
// challenge.  --verify re-derives the challenges from the statement AND the serialised proof's own round messages (zk_fs_round_challenges).
// --repeat N proves the same statement N more times after the reference's single (cold: first-use allocations, window tables)
// timed loop and prints the best of them as "Warm proof time: ..." -- the transcript kept is the first run's.
// --full-tables C: digit tables of every layer's generators (zk_gens_prepare_full, window C bits; 83 GB for the 18M MLP at C = 12), built with the
// commitments (setup); the openings' MSMs then need no buckets.
// --gpus N (N > 1): this process starts N copies of itself, one per GPU (ZKDL_RANK / ZKDL_WORLD / ZKDL_DEVICE in their environment), BEFORE it
// touches the GPU; every rank proves the statement on its own device with its own generator seed (independent proof batches: north_star's
// multi-GPU layout), and the ranks' transcript digests are summed by ONE RCCL all-reduce over xGMI (zk_dist_*: no torch.distributed); rank 0
// prints "Aggregate over N ranks: ...".  The 128-byte RCCL id travels from rank 0 to the others through a file.  --dist runs the same
// collective path with a single rank (what a 1-GPU box can exercise).
#include <torch/script.h>
#include <torch/torch.h>

#include <algorithm>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <thread>
#include <cstring>
#include <fstream>
#include <iterator>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "zkdl.hpp"

using namespace std;

// device copy of a float tensor's contents (zkFC::from_float_gpu_ptr wants a device pointer)
struct DeviceFloats {
  float* p = nullptr;
  explicit DeviceFloats(const torch::Tensor& t) {
    auto c = t.to(torch::kCPU).to(torch::kFloat32).contiguous();
    void* d = nullptr;
    zkdl::check(zk_malloc(zkdl::ctx(), (size_t)c.numel() * sizeof(float), &d));
    zkdl::check(zk_memcpy_h2d(zkdl::ctx(), d, c.data_ptr<float>(), (size_t)c.numel() * sizeof(float)));
    p = static_cast<float*>(d);
  }
  ~DeviceFloats() { zk_free(zkdl::ctx_slot(), p); }
};

FrTensor fcnn_inference(const FrTensor& X, const vector<zkFC>& fcs, vector<zkReLU>& relus, vector<FrTensor>& Z_vec, vector<FrTensor>& A_vec) {   // demo.cu:23-38
  if (fcs.size() != relus.size() + 1) throw std::runtime_error("Incompatible number of layers");
  uint num_layer = fcs.size();
  for (uint i = 0; i < num_layer - 1; ++i) {
    const auto& A = (i == 0) ? X : A_vec[i - 1];
    Z_vec.push_back(fcs[i](A));
    A_vec.push_back(relus[i](Z_vec[i]));
  }
  return fcs[num_layer - 1](A_vec[num_layer - 2]);
}

torch::Tensor load_tensor(const string& path) {   // demo.cu:40-46
  torch::Tensor tensor;
  torch::load(tensor, path);
  return tensor;
}

// demo.cu:48-95, generalised: Linear layers WITH bias are accepted.  The protocol itself has no bias term
// (zkFC proves Z = X W, zkfc.cu:117-145), so the loader folds the biases into the matrices: every activation carries one extra
// "ones" feature (1.0, i.e. 2^16 after quantisation), every weight matrix one extra input row holding the bias and -- except the
// last -- one extra output column [0 .. 0 1]^T that passes the ones feature on (Z = 2^32 there, which the ReLU rescales back to
// 2^16).  The augmented bias-free network computes exactly the biased one, inside the power-of-two padding wherever in + 1 <= in_p.
// The activation between two Linear layers (loader generalisation): nn.ReLU (the reference's only one, demo.cu:61-72 skips
// it by name), nn.LeakyReLU whose negative slope is a power of two 2^-s, nn.Identity, or nothing at all (= identity: the layers still
// need the requantisation).  {kind, shift} as zk_relu_set_activation takes them.
struct Act {
  uint kind = 0, shift = 0;
};
static uint g_full_tables = 0;   // --full-tables C
static Act activation_of(const torch::jit::script::Module& child) {
  const string type = child.type()->name() ? child.type()->name()->qualifiedName() : string();
  Act a;
  if (type.find("LeakyReLU") != string::npos) {
    double slope = 0.01;   // nn.LeakyReLU's default
    for (const auto* node : child.get_method("forward").graph()->nodes())
      if (string(node->kind().toQualString()) == "aten::leaky_relu") {
        auto v = torch::jit::toIValue(node->input(1));
        if (v && v->isDouble()) slope = v->toDouble();
        else if (v && v->isInt()) slope = (double)v->toInt();
      }
    uint s = 0;
    while (s < 32 && std::ldexp(1.0, -(int)s) != slope) s++;
    if (s >= 32) throw std::runtime_error("LeakyReLU: negative slope " + to_string(slope) + " is not a power of two 2^-s, s <= 31");
    a.kind = 1;
    a.shift = s;
  } else if (type.find("Identity") != string::npos) {
    a.kind = 1;
    a.shift = 0;
  } else if (type.find("ReLU") == string::npos) {
    throw std::runtime_error("unsupported activation module: " + type);
  }
  return a;
}

vector<zkFC> load_model(const string& model_path, vector<unique_ptr<Commitment>>& generators, bool& has_bias, vector<Act>& acts) {
  vector<zkFC> fcs;
  torch::jit::script::Module m;
  try {
    m = torch::jit::load(model_path, torch::kCPU);
  } catch (const c10::Error& e) {
    std::cerr << "Error loading the model\n";
    exit(-1);
  }
  vector<torch::Tensor> Ws, Bs;
  has_bias = false;
  for (int i = 0;; ++i) {
    if (!m.hasattr(to_string(i))) break;
    auto child = m.attr(to_string(i)).toModule();
    if (!child.hasattr("weight")) {   // an activation: belongs behind the Linear layer before it
      if (!Ws.empty() && acts.size() + 1 == Ws.size()) acts.push_back(activation_of(child));
      else throw std::runtime_error("Incompatible number of layers");
      continue;
    }
    if (acts.size() + 1 == Ws.size()) {   // two Linear layers in a row: identity (requantisation only)
      Act a;
      a.kind = 1;
      acts.push_back(a);
    }
    Ws.push_back(child.attr("weight").toTensor().t().to(torch::kCPU).to(torch::kFloat32).contiguous());   // [in, out]
    torch::Tensor b;
    if (child.hasattr("bias") && child.attr("bias").isTensor()) {
      b = child.attr("bias").toTensor().to(torch::kCPU).to(torch::kFloat32).contiguous();
      has_bias = true;
    }
    Bs.push_back(b);
  }
  uint parameter_count = 0;
  for (size_t l = 0; l < Ws.size(); ++l) {
    auto weight_tensor = Ws[l];
    parameter_count += weight_tensor.numel() + (Bs[l].defined() ? Bs[l].numel() : 0);
    if (has_bias) {
      const int64_t in = weight_tensor.size(0), out = weight_tensor.size(1);
      const bool last = l + 1 == Ws.size();
      auto aug = torch::zeros({in + 1, out + (last ? 0 : 1)}, torch::kFloat32);
      aug.narrow(0, 0, in).narrow(1, 0, out).copy_(weight_tensor);
      if (Bs[l].defined()) aug.narrow(0, in, 1).narrow(1, 0, out).copy_(Bs[l].view({1, out}));
      if (!last) aug.index_put_({in, out}, 1.0f);
      weight_tensor = aug.contiguous();
    }
    int in_dim = weight_tensor.sizes()[0], out_dim = weight_tensor.sizes()[1];
    DeviceFloats w(weight_tensor);
    generators.emplace_back(new Commitment(1U << ((ceilLog2(in_dim * out_dim) + 1) / 2), G1Jacobian_generator));   // demo.cu:81
    *generators.back() *= FrTensor::random(generators.back()->size);                                               // demo.cu:82
    fcs.push_back(zkFC::from_float_gpu_ptr(in_dim, out_dim, w.p, *generators.back()));
    if (g_full_tables) generators.back()->prepare_full(g_full_tables);
    if (fcs.size() > 1 && fcs[fcs.size() - 2].outputSize != fcs[fcs.size() - 1].inputSize) throw std::runtime_error("Incompatible layer sizes");
  }
  cout << "Total number of parameters: " << parameter_count << endl;
  return fcs;
}

// --gpus N: the parent of the N ranks.  It never touches the GPU (the ranks are exec'ed copies of this program).
static int spawn_ranks(int n, char* argv[]) {
  char idfile[] = "/tmp/zkdl_rccl_id_XXXXXX";
  int fd = mkstemp(idfile);
  if (fd >= 0) close(fd);
  unlink(idfile);   // rank 0 creates it (write to .tmp, rename): the others wait for it to appear
  vector<pid_t> pids;
  for (int r = 0; r < n; r++) {
    pid_t pid = fork();
    if (pid == 0) {
      setenv("ZKDL_RANK", to_string(r).c_str(), 1);
      setenv("ZKDL_WORLD", to_string(n).c_str(), 1);
      if (!getenv("ZKDL_DIST_SAME_DEVICE")) setenv("ZKDL_DEVICE", to_string(r).c_str(), 1);
      setenv("ZKDL_DIST_ID_FILE", idfile, 1);
      setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
      execv("/proc/self/exe", argv);
      _exit(127);
    }
    pids.push_back(pid);
  }
  int rc = 0;
  for (pid_t pid : pids) {
    int st = 0;
    waitpid(pid, &st, 0);
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128;
    if (code && !rc) rc = code;
  }
  unlink(idfile);
  return rc;
}

int main(int argc, char* argv[]) {
  if (argc < 3) {
    cerr << "usage: demo traced_model.pt sample_input.pt [--seed S]\n";
    return 2;
  }
  int gpus = 1;
  bool dist = false;
  for (int i = 3; i < argc; i++) {
    if (string(argv[i]) == "--gpus" && i + 1 < argc) gpus = atoi(argv[i + 1]);
    if (string(argv[i]) == "--dist") dist = true;
    if (string(argv[i]) == "--full-tables" && i + 1 < argc) g_full_tables = (uint)atoi(argv[i + 1]);
  }
  const char* rank_env = getenv("ZKDL_RANK");
  if (gpus > 1 && !rank_env) return spawn_ranks(gpus, argv);
  const int rank = rank_env ? atoi(rank_env) : 0, world = rank_env ? atoi(getenv("ZKDL_WORLD") ? getenv("ZKDL_WORLD") : "1") : 1;
  dist = dist || world > 1;
  string dump_dir;
  vector<Fr_t> challenges;
  bool use_fs = false, sound = false, do_verify = false;
  uint8_t fs_seed[32] = {};
  int repeat = 0, pipeline = 0, pipe_warmup = 5, inflight = 8, g1_batch = 3;
  bool pipe_derive = true;
  double spinup_ms = 400.0;
  uint64_t chal_seed = 0x5EEDC4A100000000ull;   // bench.py: Workload.challenges draws step k's vectors from seed (this + k)
  for (int i = 3; i < argc; i++) {
    if (string(argv[i]) == "--repeat" && i + 1 < argc) repeat = atoi(argv[i + 1]);
    if (string(argv[i]) == "--pipeline" && i + 1 < argc) pipeline = atoi(argv[i + 1]);
    if (string(argv[i]) == "--warmup" && i + 1 < argc) pipe_warmup = atoi(argv[i + 1]);
    if (string(argv[i]) == "--inflight" && i + 1 < argc) inflight = atoi(argv[i + 1]);
    if (string(argv[i]) == "--g1-batch" && i + 1 < argc) g1_batch = atoi(argv[i + 1]);
    if (string(argv[i]) == "--no-derive") pipe_derive = false;
    if (string(argv[i]) == "--spinup-ms" && i + 1 < argc) spinup_ms = atof(argv[i + 1]);
    if (string(argv[i]) == "--chal-seed" && i + 1 < argc) chal_seed = strtoull(argv[i + 1], nullptr, 0);
    if (string(argv[i]) == "--seed" && i + 1 < argc) zkdl::seed(strtoull(argv[i + 1], nullptr, 0) + (uint64_t)rank);   // every rank its own generators
    if (string(argv[i]) == "--dump" && i + 1 < argc) dump_dir = argv[i + 1];
    if (string(argv[i]) == "--fs") use_fs = true;
    if (string(argv[i]) == "--sound") use_fs = sound = true;
    if (string(argv[i]) == "--verify") do_verify = true;
  }
  zkdl::challenge_log() = &challenges;
  vector<unique_ptr<Commitment>> generators;
  bool has_bias = false;
  vector<Act> acts;
  vector<zkFC> fcs = load_model(argv[1], generators, has_bias, acts);
  vector<zkReLU> relus(fcs.size() - 1);
  if (acts.size() + 1 < fcs.size() || acts.size() > fcs.size()) throw std::runtime_error("Incompatible number of layers");
  for (size_t i = 0; i + 1 < fcs.size(); i++) {
    // (with folded biases the constant "ones" feature is positive, so it passes every one of these activations unchanged)
    if (acts[i].kind) relus[i].set_activation(acts[i].kind, acts[i].shift);
  }
  vector<FrTensor> Z_vec, A_vec;
  Z_vec.reserve(fcs.size());
  A_vec.reserve(fcs.size());

  torch::Tensor sample_input = load_tensor(argv[2]).to(torch::kCPU).to(torch::kFloat32);
  if (has_bias) sample_input = torch::cat({sample_input, torch::ones({sample_input.size(0), 1}, torch::kFloat32)}, 1);   // the "ones" feature
  int batch_size = sample_input.size(0), input_dim = sample_input.size(1);
  DeviceFloats xin(sample_input);
  auto X = zkFC::load_float_gpu_input(batch_size, input_dim, xin.p);

  auto Y_hat = fcnn_inference(X.mont(), fcs, relus, Z_vec, A_vec).unmont();   // demo.cu:119
  ofstream outfile(rank == 0 ? string("demo.out") : "demo.rank" + to_string(rank) + ".out");
  outfile << Y_hat << endl;
  outfile.close();

  auto num_layer = fcs.size();
  if (use_fs) {  // statement digest: the two input files and every layer's commitment points
    auto slurp = [](const char* path) { ifstream f(path, ios::binary); return string((istreambuf_iterator<char>(f)), istreambuf_iterator<char>()); };
    string st = "zkdl-amd/demo/v1";
    uint8_t d[32];
    for (const char* path : {argv[1], argv[2]}) {
      string bytes = slurp(path);
      zk_sha256(bytes.data(), bytes.size(), d);
      st.append(reinterpret_cast<char*>(d), 32);
    }
    for (auto& fc : fcs) {
      vector<G1Jacobian_t> c(fc.commitment_size());
      zkdl::check(zk_memcpy_d2h(zkdl::ctx(), c.data(), fc.commitment_points(), c.size() * sizeof(G1Jacobian_t)));
      zk_sha256(c.data(), c.size() * sizeof(G1Jacobian_t), d);
      st.append(reinterpret_cast<char*>(d), 32);
    }
    zk_sha256(st.data(), st.size(), d);
    zkdl::fs_begin(d);
    memcpy(fs_seed, d, 32);
  }
  auto prove_calls = [&]() {   // demo.cu:124-138: the prove calls themselves (they append to zkdl::transcript())
    if (sound) {
      fcs[num_layer - 1].prove_fs(A_vec[num_layer - 2], Y_hat, *generators[num_layer - 1], fs_seed, num_layer - 1);
      for (int i = num_layer - 2; i >= 0; --i) {
        relus[i].prove_fs(Z_vec[i], A_vec[i], fs_seed, i);
        fcs[i].prove_fs((i > 0) ? A_vec[i - 1] : X, Z_vec[i], *generators[i], fs_seed, i);
      }
      return;
    }
    if (use_fs) zkdl::fs_call(num_layer - 1, 0);
    fcs[num_layer - 1].prove(A_vec[num_layer - 2], Y_hat, *generators[num_layer - 1]);
    for (int i = num_layer - 2; i >= 0; --i) {
      if (use_fs) zkdl::fs_call(i, 1);
      relus[i].prove(Z_vec[i], A_vec[i]);
      FrTensor& A_ = (i > 0) ? A_vec[i - 1] : X;
      if (use_fs) zkdl::fs_call(i, 0);
      fcs[i].prove(A_, Z_vec[i], *generators[i]);
    }
  };
  auto prove_all = [&]() {
    prove_calls();
    zk_transcript_flush(zkdl::transcript());   // the deferred MSM batch belongs to the timed region
    zk_ctx_sync(zkdl::ctx());
  };
  Timer timer;
  zk_ctx_sync(zkdl::ctx());
  timer.start();
  prove_all();
  timer.stop();

  size_t n_fr = 0, n_g1 = 0;
  zk_transcript_counts(zkdl::transcript(), &n_fr, &n_g1);
  Fr_t digest;
  zk_transcript_digest(zkdl::transcript(), &digest);
  cout << "Proof time: " << timer.getTotalTime() / batch_size << " seconds per data point." << endl;
  cout << "Current HIP status: " << 0 << endl;
  cout << "Transcript: " << n_fr << " Fr + " << n_g1 << " G1 values, digest " << digest << endl;
  // serialised now: --repeat proves again into the same transcript object
  vector<unsigned char> first_blob;
  if (do_verify || !dump_dir.empty()) {
    size_t len = 0;
    zk_transcript_serialize(zkdl::transcript(), nullptr, 0, &len);
    first_blob.resize(len);
    zkdl::check(zk_transcript_serialize(zkdl::transcript(), first_blob.data(), first_blob.size(), &len));
  }
  if (sound && (do_verify || !dump_dir.empty())) {
    // the verifier's derivation: evaluation points from the statement digest, round challenges from the proof's own round messages
    challenges.clear();
    const Fr_t* fr = reinterpret_cast<const Fr_t*>(first_blob.data() + 24);
    const uint B_p = 1u << ceilLog2(batch_size);
    size_t off = 0;
    auto derived = [&](uint layer, uint kind, uint vec, uint len) {
      vector<Fr_t> v(len);
      zkdl::check(zk_fs_derive(fs_seed, layer, kind, vec, len, v.data()));
      challenges.insert(challenges.end(), v.begin(), v.end());
    };
    auto rounds = [&](uint layer, uint kind, uint which, size_t at, uint k) {
      vector<Fr_t> v(k);
      if (at + 3 * (size_t)k > n_fr) throw std::runtime_error("transcript does not have the shape of this model's proof");
      zkdl::check(zk_fs_round_challenges(fs_seed, layer, kind, which, fr + at, k, v.data()));
      challenges.insert(challenges.end(), v.begin(), v.end());
    };
    auto fc_ch = [&](uint i) {
      const uint k_in = ceilLog2(fcs[i].inputSize), k_out = ceilLog2(fcs[i].outputSize);
      derived(i, 0, 0, ceilLog2(B_p));
      rounds(i, 0, 0, off, k_in);
      derived(i, 0, 2, k_out);
      off += 3 * k_in + 4;
    };
    fc_ch(num_layer - 1);
    for (int i = num_layer - 2; i >= 0; --i) {
      const uint k = ceilLog2(B_p * fcs[i].outputSize);
      const size_t o_mag = off, o_rem = off + 3 * (k + 5) + 1 + 32, o_hp = o_rem + 3 * (k + 4) + 1 + 16;
      derived(i, 1, 0, k + 5);
      rounds(i, 1, 1, o_mag, k + 5);
      derived(i, 1, 2, k + 4);
      rounds(i, 1, 2, o_rem, k + 4);
      derived(i, 1, 4, k);
      derived(i, 1, 5, k);
      rounds(i, 1, 3, o_hp, k);
      off = o_hp + 3 * k + 2;
      if (relus[i].leaky()) {   // the extra Hadamard sumcheck of a LeakyReLU / identity layer
        derived(i, 1, 7, k);
        rounds(i, 1, 4, off, k);
        off += 3 * k + 2;
      }
      fc_ch(i);
    }
  }
  const vector<Fr_t> first_challenges = challenges;
  if (repeat > 0) {
    double best = 1e30;
    for (int it = 0; it < repeat; it++) {
      zk_transcript_reset(zkdl::transcript());
      challenges.clear();
      Timer t2;
      zk_ctx_sync(zkdl::ctx());
      t2.start();
      prove_all();
      t2.stop();
      best = std::min(best, t2.getTotalTime());
    }
    Fr_t d2;
    zk_transcript_digest(zkdl::transcript(), &d2);
    cout << "Warm proof time: " << best / batch_size << " seconds per data point (best of " << repeat << "; " << best * 1e3 << " ms per batch-" << batch_size << " proof)" << endl;
    if (use_fs && memcmp(&d2, &digest, sizeof digest) != 0) {   // hash-derived challenges: the same statement must give the same proof
      cout << "Repeat: transcript digest differs from the first run" << endl;
      return 1;
    }
    challenges = first_challenges;
  }
  zk_dist* dist_h = nullptr;
  auto dist_connect = [&]() {   // rank 0 makes the RCCL id; the 128 bytes travel through a file
    uint8_t id[128];
    const char* idfile = getenv("ZKDL_DIST_ID_FILE");
    if (rank == 0) {
      zkdl::check(zk_dist_unique_id(id));
      if (idfile) {
        const string tmp = string(idfile) + ".tmp";
        ofstream f(tmp, ios::binary);
        f.write(reinterpret_cast<const char*>(id), 128);
        f.close();
        rename(tmp.c_str(), idfile);
      }
    } else {
      if (!idfile) throw std::runtime_error("ZKDL_DIST_ID_FILE is not set");
      for (int tries = 0;; tries++) {
        ifstream f(idfile, ios::binary);
        if (f && f.read(reinterpret_cast<char*>(id), 128)) break;
        if (tries > 3000) throw std::runtime_error("timed out waiting for rank 0's RCCL id");
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
      }
    }
    zkdl::check(zk_dist_init(zkdl::ctx(), rank, world, id, &dist_h));
  };
  if (pipeline > 0) {
    // ---- the throughput loop (bench.py's timed region, in C++): K proofs of this statement with fresh challenges each, `inflight` transcripts
    // rotating, `g1_batch` sealed proofs per G1 batch, derived round commitments, the run's last G1 batch in the short form; the digests are
    // summed on the device and -- with --gpus N / --dist -- all-reduced ONCE over RCCL.  Step k draws its challenges from seed (chal_seed + k) in
    // the reference's order (zkfc.cu:135-137, zkrelu.cu:85-98), exactly as bench.py does: same statement + same seeds = same run aggregate.
    if (dist) dist_connect();
    zkdl::challenge_log() = nullptr;
    const size_t cap = (size_t)std::max({pipeline, pipe_warmup, 64});
    zkdl::ProofPipeline pipe(cap, (uint)inflight, (uint)g1_batch, pipe_derive && !sound);
    auto step = [&](int64_t k) {
      if (!use_fs) zkdl::seed(chal_seed + (uint64_t)(k & 0xffffffff));
      pipe.submit(prove_calls);
    };
    auto t_spin = std::chrono::steady_clock::now();
    int64_t n_spin = 0;
    while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_spin).count() < spinup_ms) {   // clocks and pools up (untimed)
      if (pipe.submitted() == pipe.capacity()) pipe.rewind(cap);
      step(-100000 - n_spin++);
      if (n_spin % 8 == 0) zk_ctx_sync(zkdl::ctx());
    }
    pipe.rewind((size_t)std::max(pipe_warmup, 1));
    for (int k = 0; k < pipe_warmup; k++) step(-1 - k);   // its last G1 batch runs in the drain form: pools hold blocks of both shapes
    pipe.rewind((size_t)pipeline);
    if (dist_h) {   // warm the communicator too
      uint64_t* w = zkdl::dalloc<uint64_t>(8);
      zkdl::check(zk_dist_sum_rows(zkdl::ctx(), pipe.rows(), 1, w));
      zkdl::check(zk_dist_allreduce_u64(dist_h, w, 8));
      zk_free(zkdl::ctx(), w);
    }
    zk_ctx_sync(zkdl::ctx());
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < pipeline; k++) step(k);
    const double enq_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    Fr_t agg = pipe.finish(dist_h);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rank == 0) {
      cout << "Pipelined proof time: " << ms / 1e3 / pipeline / batch_size << " seconds per data point (" << pipeline << " proofs, " << ms / pipeline << " ms per batch-"
           << batch_size << " proof; " << inflight << " in flight, " << g1_batch << " per G1 batch, host enqueue " << enq_ms / pipeline << " ms per proof)" << endl;
      cout << "Run aggregate" << (dist_h ? " over " + to_string(zk_dist_world_size(dist_h)) + " ranks (RCCL all-reduce)" : string()) << ": " << agg << endl;
    }
    zkdl::challenge_log() = &challenges;
    challenges = first_challenges;
    if (dist_h) {
      zk_dist_destroy(dist_h);
      cout.flush();
      zk_ctx_sync(zkdl::ctx());
      _exit(0);
    }
    dist = false;
  }
  if (dist) {   // the multi-GPU aggregate: ONE RCCL all-reduce of the ranks' transcript digests (8 x uint64 lanes), no torch
    dist_connect();
    zk_dist* d = dist_h;
    uint64_t* lanes_d = zkdl::dalloc<uint64_t>(8);
    zkdl::check(zk_transcript_digest_dev(zkdl::transcript(), lanes_d));
    zkdl::check(zk_dist_allreduce_u64(d, lanes_d, 8));
    uint64_t lanes[8];
    zkdl::check(zk_memcpy_d2h(zkdl::ctx(), lanes, lanes_d, sizeof lanes));
    Fr_t agg;
    zk_dist_lanes_to_fr(lanes, &agg);
    if (rank == 0) cout << "Aggregate over " << zk_dist_world_size(d) << " ranks (RCCL all-reduce of the transcript digests): " << agg << endl;
    zk_dist_destroy(d);
    zk_free(zkdl::ctx(), lanes_d);
  }
  if (do_verify) {  // the verifier's pass: public data + the serialised proof only
    const vector<unsigned char>& buf = first_blob;
    vector<Fr_t> fr(n_fr);
    vector<G1Affine_t> g1(n_g1);
    vector<uint8_t> inf(n_g1);
    memcpy(fr.data(), buf.data() + 24, n_fr * sizeof(Fr_t));
    memcpy(g1.data(), buf.data() + 24 + n_fr * sizeof(Fr_t), n_g1 * sizeof(G1Affine_t));
    memcpy(inf.data(), buf.data() + 24 + n_fr * sizeof(Fr_t) + n_g1 * sizeof(G1Affine_t), n_g1);
    const uint B_p = 1u << ceilLog2(batch_size);
    size_t of = 0, og = 0, oc = 0;
    auto take = [&](uint n) { vector<Fr_t> v(challenges.begin() + oc, challenges.begin() + oc + n); oc += n; return v; };
    string verdict;
    auto verify_fc = [&](int i) {
      const zkFC& fc = fcs[i];
      auto u_bs = take(ceilLog2(B_p)), u_in = take(ceilLog2(fc.inputSize)), u_out = take(ceilLog2(fc.outputSize));
      const size_t nf = 3 * ceilLog2(fc.inputSize) + 4, ng = 3 * ceilLog2(generators[i]->size) + 2;
      Fr_t ez, ex;
      const bool last = i == (int)num_layer - 1, first = i == 0;
      if (last) { auto uz = concatenate<Fr_t>({u_out, u_bs}); ez = Y_hat(uz); }
      if (first) { auto ux = concatenate<Fr_t>({u_in, u_bs}); ex = X(ux); }
      string msg = fc.verify(B_p, *generators[i], last, u_bs, u_in, u_out, fr.data() + of, nf, g1.data() + og, inf.data() + og, ng, last ? &ez : nullptr, first ? &ex : nullptr);
      of += nf;
      og += ng;
      if (!msg.empty() && verdict.empty()) verdict = "zkFC layer " + to_string(i) + ": " + msg;
    };
    verify_fc(num_layer - 1);
    for (int i = num_layer - 2; i >= 0; --i) {
      const uint n = B_p * fcs[i].outputSize, k = ceilLog2(n);
      vector<vector<Fr_t>> ch;
      for (uint l : {k + 5, k + 5, k + 4, k + 4, k, k, k}) ch.push_back(take(l));
      size_t nf = 3 * (k + 5) + 1 + 32 + 3 * (k + 4) + 1 + 16 + 3 * k + 2;
      string msg;
      if (relus[i].leaky()) {
        ch.push_back(take(k));
        ch.push_back(take(k));
        nf += 3 * k + 2;
        msg = zkReLU::verify_leaky(n, ch, fr.data() + of, nf);
      } else {
        msg = zkReLU::verify(n, ch, fr.data() + of, nf);
      }
      of += nf;
      if (!msg.empty() && verdict.empty()) verdict = "zkReLU layer " + to_string(i) + ": " + msg;
      verify_fc(i);
    }
    if (of != n_fr || og != n_g1 || oc != challenges.size()) verdict = "transcript does not have the shape of this model's proof";
    cout << "Verification: " << (verdict.empty() ? "OK" : verdict) << endl;
    if (!verdict.empty()) return 1;
  }
  if (!dump_dir.empty()) {
    auto put = [&](const string& name, const void* p, size_t bytes) {
      ofstream f(dump_dir + "/" + name, ios::binary);
      f.write(static_cast<const char*>(p), (streamsize)bytes);
    };
    for (size_t i = 0; i < generators.size(); i++) {
      vector<G1Jacobian_t> g(generators[i]->size);
      zkdl::check(zk_memcpy_d2h(zkdl::ctx(), g.data(), generators[i]->data(), g.size() * sizeof(G1Jacobian_t)));
      put("gens_" + to_string(i) + ".bin", g.data(), g.size() * sizeof(G1Jacobian_t));
    }
    put("challenges.bin", challenges.data(), challenges.size() * sizeof(Fr_t));
    put("transcript.bin", first_blob.data(), first_blob.size());
  }
  if (dist) {
    // this process holds two HIP runtimes (LibTorch's bundled one and ROCm's, which libzkdl_amd.so and librccl.so.1 use); RCCL's and their
    // static destructors do not agree on an order at exit.  Everything is written: leave without running them.
    cout.flush();
    cerr.flush();
    zk_ctx_sync(zkdl::ctx());
    _exit(0);
  }
  return 0;
}

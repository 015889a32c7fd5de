"""The C++ mirror of the reference API (zkdl_amd/host/zkdl.hpp) and the demo driver, on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "zkdl_amd", "host")


def test_cpp_class_api():
    exe = os.path.join(HOST, "test_host_api")
    if not os.path.exists(exe):
        pytest.skip("zkdl_amd/host/test_host_api not built")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host api ok" in out.stdout


def test_demo_driver_matches_float_model_and_oracle(tmp_path, oracle):
    """model.py -> ./demo: demo.out (Y_hat, un-Montgomery'd, padded) equals the oracle's quantised
    forward bit for bit and de-quantises to the float network's output."""
    exe = os.path.join(HOST, "demo")
    if not os.path.exists(exe):
        pytest.skip("zkdl_amd/host/demo not built (needs LibTorch)")
    import torch
    import bigint_model as bm
    d = str(tmp_path)
    subprocess.run([sys.executable, "-m", "zkdl_amd.model", "--dims", "6,5,4", "--batch", "3", "--seed", "7", "--out-dir", d], check=True, cwd=ROOT)
    run = subprocess.run([exe, os.path.join(d, "traced_model.pt"), os.path.join(d, "sample_input.pt"), "--seed", "5"], capture_output=True, text=True, cwd=d, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "Total number of parameters: 50" in run.stdout and "seconds per data point." in run.stdout
    lines = open(os.path.join(d, "demo.out")).read().strip().strip("[]").split("\n")
    vals = [int(x, 16) for x in lines]
    assert len(vals) == 4 * 4                                    # B_p x out_p
    m = torch.jit.load(os.path.join(d, "traced_model.pt"))
    W = [p.detach().numpy().T.copy() for p in m.parameters()]    # [in, out]
    x = list(torch.jit.load(os.path.join(d, "sample_input.pt")).parameters())[0].detach().numpy()
    want = np.maximum(x.astype(np.float64) @ W[0], 0) @ W[1]
    for b in range(3):
        for o in range(4):
            v = vals[b * 4 + o]
            v = v - bm.R_MOD if v > bm.R_MOD // 2 else v
            assert abs(v / 2.0 ** 32 - want[b, o]) < 2e-3
    # bit-exact against the oracle's quantised forward
    Wq = [oracle.fr_mont(oracle.float_to_fr(W[0], 8, 8)), oracle.fr_mont(oracle.float_to_fr(W[1], 8, 4))]
    X = oracle.fr_mont(oracle.float_to_fr(x, 4, 8))
    A, *_ = oracle.relu_forward(oracle.fc_forward(X, Wq[0], 4, 8, 8))
    Y = oracle.fr_unmont(oracle.fc_forward(A, Wq[1], 4, 8, 4))
    assert vals == [bm.limbs_to_int(r) for r in Y]
